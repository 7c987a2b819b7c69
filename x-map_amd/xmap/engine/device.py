"""Device engine: drives the three stages of the hot path through the C ABI (include/xmap_hip.h).

torch is used only as the container for HBM buffers and for the current HIP stream; every
computation on the path is a kernel of libxmap_hip.so.  All results stay resident in HBM as
torch tensors until a caller asks for host copies.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import hipabi as abi

lib = abi.lib
vp, i32, i64, check = abi.vp, abi.i32, abi.i64, abi.check


def _stream(dev):
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


# Stage A: partner bound of one hash partition of a row in the 1024-slot LDS table (75 % load, like the 96 / 192 / 384 of the
# smaller table classes) and the least rater count of a heavy row (profiles/tools/tune_a.py at BASELINE configs[1]: pair kernels
# 1.28 ms at 640 / 1024, 1.20 ms at 768 / 2048)
SLOT_TARGET = 768
CH_MIN = 2048


class DeviceRatings(object):
    """trainRDD in index space, resident in HBM: CSR by user (trainRDD / profile order).  The CSC by item is
    derived on the device at the start of every stage-A pass (Engine.build_csc); only the id dictionary and the
    H2D upload are one-off host work."""

    def __init__(self, user_ptr, item, rating, time, n_items, attrs, device="cuda:0", rating64=False):
        """rating64: also keep the ratings as fp64 (RecommenderSim runs over AlterEgo rows, whose ratings are np.float64 means,
        reference core/generator.py:123-138 -> core/recommenderSim.py:64-133; the three stages of the path read float32)"""
        self.device = torch.device(device)
        user_ptr = np.ascontiguousarray(user_ptr, np.int64)
        item = np.ascontiguousarray(item, np.int32)
        r64 = np.ascontiguousarray(rating, np.float64) if rating64 else None
        rating = np.ascontiguousarray(rating, np.float32)
        time = np.ascontiguousarray(time, np.int64)
        self.n_users = len(user_ptr) - 1
        self.n_items = int(n_items)
        self.nnz = int(user_ptr[-1])
        # contributions of the "tri" formulation (every unordered pair of a profile once): a property of the profile
        # lengths, so the engine sizes its pair buffers and unit arrays without asking the device
        d = np.diff(user_ptr)
        self.half_contrib = int((d * (d - 1) // 2).sum())
        if self.nnz >= 2 ** 31 - 1:
            raise ValueError("nnz must fit int32")
        if self.nnz and (item.min() < 0 or item.max() >= self.n_items):
            raise ValueError("item index out of range")
        prefix_cls, suffix_cls, contains_mask, flags = attrs
        d = self.device
        t = torch.from_numpy
        self.user_ptr = t(user_ptr).to(d)
        self.user_item = t(item).to(d)
        self.user_rating = t(rating).to(d)
        self.user_rating64 = t(r64).to(d) if r64 is not None else None
        self.user_time = t(time).to(d)
        self.item_ptr = torch.zeros(self.n_items + 1, dtype=torch.int64, device=d)
        self.item_user = torch.zeros(max(self.nnz, 1), dtype=torch.int32, device=d)
        self.item_rating = torch.zeros(max(self.nnz, 1), dtype=torch.float32, device=d)
        self.csc_ready = False
        self.prefix_cls = t(np.ascontiguousarray(prefix_cls, np.int32)).to(d)
        self.suffix_cls = t(np.ascontiguousarray(suffix_cls, np.int32)).to(d)
        self.contains_mask = t(np.ascontiguousarray(contains_mask, np.uint32).view(np.int32)).to(d)
        self.flags = t(np.ascontiguousarray(flags, np.uint8)).to(d)
        self.c = abi.Ratings(self.n_users, self.n_items, self.nnz,
                             self.user_ptr.data_ptr(), self.user_item.data_ptr(), self.user_rating.data_ptr(),
                             self.user_time.data_ptr(), self.item_ptr.data_ptr(), self.item_user.data_ptr(),
                             self.item_rating.data_ptr(), self.prefix_cls.data_ptr(), self.suffix_cls.data_ptr(),
                             self.contains_mask.data_ptr(), self.flags.data_ptr())

    def bytes_resident(self):
        return sum(x.numel() * x.element_size() for x in (
            self.user_ptr, self.user_item, self.user_rating, self.user_time, self.item_ptr,
            self.item_user, self.item_rating))


class SimResult(object):
    """Stage-A output in HBM: CSR (row_ptr, col, sim fp64, mutu, nij) + item/user info."""
    pass


class ExtResult(object):
    pass


class GenResult(object):
    pass


class _Timed(object):
    """HIP-event bracket around one C-ABI call on the engine's stream (only when profiling is on)."""

    def __init__(self, eng, name):
        self.eng, self.name = eng, name

    def __enter__(self):
        if self.eng.timers is not None:
            self.a = torch.cuda.Event(enable_timing=True)
            self.b = torch.cuda.Event(enable_timing=True)
            self.a.record(torch.cuda.current_stream(self.eng.dev))
        return self

    def __exit__(self, *exc):
        if self.eng.timers is not None:
            self.b.record(torch.cuda.current_stream(self.eng.dev))
            self.eng.timers.setdefault(self.name, []).append((self.a, self.b))
        return False


class Engine(object):
    def __init__(self, ratings):
        self.R = ratings
        self.dev = ratings.device
        self.timers = None  # set to {} to collect per-call HIP-event timings
        self._scratch = {}  # persistent zero-filled accumulator rows (the path kernels leave them zeroed)

    def _zero_scratch(self, name, numel, dtype):
        """Zero-filled scratch that kernels return zeroed: allocated (and cleared) once, reused across passes."""
        t = self._scratch.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.zeros(numel, dtype=dtype, device=self.dev)
            self._scratch[name] = t
        return t

    def hbm_available(self, *own):
        """bytes this process could allocate now: free on the device + unused blocks of torch's caching allocator + the
        scratch buffers named in `own` (about to be re-used or replaced)"""
        free, _ = torch.cuda.mem_get_info(self.dev)
        cached = torch.cuda.memory_reserved(self.dev) - torch.cuda.memory_allocated(self.dev)
        held = sum(t.numel() * t.element_size() for n, t in self._scratch.items() if n in own)
        return int(free + max(cached, 0) + held)

    def _drop_scratch(self, *names):
        """forget persistent zero-filled scratch (after a failed pass its rows may hold partial sums)"""
        for n in names:
            self._scratch.pop(n, None)

    def timed(self, name):
        return _Timed(self, name)

    def timer_ms(self):
        """{name: [ms, ...]} of everything recorded since timers was set (synchronises)."""
        torch.cuda.synchronize(self.dev)
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in (self.timers or {}).items()}

    def _empty(self, shape, dtype):
        return torch.empty(shape, dtype=dtype, device=self.dev)

    def _out(self, shape, dtype, written):
        """buffer a kernel is about to fill completely (`written`: it will run, i.e. the input is not empty): no
        zero-fill launch in that case"""
        return self._empty(shape, dtype) if written else self._zeros(shape, dtype)

    def _zeros(self, shape, dtype):
        return torch.zeros(shape, dtype=dtype, device=self.dev)

    # ------------------------------------------------------------------ stage A
    def build_csc(self):
        """item -> raters layout of the ratings (device counting sort; part of every stage-A pass)"""
        R = self.R
        st = _stream(self.dev)
        cnt = self._empty(max(R.n_items, 1), torch.int32)
        with self.timed("build_csc"):
            check(lib.xmap_build_csc(st, i64(R.n_users), i32(R.n_items), i64(R.nnz), vp(R.user_ptr), vp(R.user_item),
                                     vp(R.user_rating), vp(cnt), vp(R.item_ptr), vp(R.item_user), vp(R.item_rating)))
        R.csc_ready = True

    def stats(self, packed=False, item_range=None):
        """A2 + A3: CSC layout, user info, item info and (packed=True: the complete-rows formulation needs them)
        the flag-packed index copies.  item_range = (lo, hi): item info of that share only (item-sharded ranks
        all-gather the rest: xmap.engine.sharded)."""
        R = self.R
        st = _stream(self.dev)
        self.build_csc()
        u_avg = self._empty(max(R.n_users, 1), torch.float64)
        u_norm = self._empty(max(R.n_users, 1), torch.float64)
        check(lib.xmap_user_stats(st, C.byref(R.c), vp(u_avg), vp(u_norm)))
        info = self._out((max(R.n_items, 1), 4), torch.float64, R.n_items > 0)
        self.norms = self._out(2 * max(R.n_items, 1), torch.float64, R.n_items > 0)
        ua_item = self._empty(max(R.nnz, 1), torch.int32) if packed else None
        ia_user = self._empty(max(R.nnz, 1), torch.int32) if packed else None
        lo, hi = (0, R.n_items) if item_range is None else (int(item_range[0]), int(item_range[1]))
        if packed:      # (the flag-packed copies are filled by the cross-check library: only the complete-rows test formulation reads them)
            abi.xcheck(abi.xlib().xmap_item_stats(st, C.byref(R.c), vp(u_avg), vp(info), vp(self.norms), vp(ua_item), vp(ia_user), i32(lo), i32(hi)))
        else:
            check(lib.xmap_item_stats(st, C.byref(R.c), vp(u_avg), vp(info), vp(self.norms), vp(ua_item), vp(ia_user), i32(lo), i32(hi)))
        return u_avg, u_norm, info, ua_item, ia_user

    def stats_partial(self):
        """user-sharded input: CSC of this rank's users, their user info, and the rank's share of the item sums
        [I][5] = (sum r, sum r^2, adjusted norm^2 as (value, error), raters) -- xmap.engine.sharded.run_step_users"""
        R = self.R
        st = _stream(self.dev)
        self.build_csc()
        u_avg = self._empty(max(R.n_users, 1), torch.float64)
        u_norm = self._empty(max(R.n_users, 1), torch.float64)
        check(lib.xmap_user_stats(st, C.byref(R.c), vp(u_avg), vp(u_norm)))
        partial = self._out((max(R.n_items, 1), 5), torch.float64, R.n_items > 0)
        check(lib.xmap_item_partials(st, C.byref(R.c), vp(u_avg), vp(partial)))
        return u_avg, u_norm, partial

    def stats_merge(self, parts):
        """parts [n_parts][I][5] (the ranks' shares in rank order) -> info [I][4]; sets self.norms"""
        I = self.R.n_items
        parts = parts.contiguous()
        info = self._out((max(I, 1), 4), torch.float64, I > 0)
        self.norms = self._out(2 * max(I, 1), torch.float64, I > 0)
        check(lib.xmap_item_merge(_stream(self.dev), i32(I), i32(int(parts.shape[0])), vp(parts), vp(info), vp(self.norms)))
        return info

    def pack_pairs(self, coo, n):
        """valid entries of a half COO -> [n][3] int64 records for the exchange of a sharded step"""
        rec = self._empty((max(n, 1), 3), torch.int64)
        cnt = C.c_int64(0)
        check(lib.xmap_sim2_pack_pairs(_stream(self.dev), i64(int(coo[0].numel())), vp(coo[0]), vp(coo[1]), vp(coo[2]), vp(coo[3]),
                                       vp(coo[4]), vp(rec), C.byref(cnt)))
        assert int(cnt.value) == n, (int(cnt.value), n)
        return rec[:n]

    def unpack_pairs(self, rec):
        n = int(rec.shape[0])
        coo = (self._empty(max(n, 1), torch.int32), self._empty(max(n, 1), torch.int32), self._empty(max(n, 1), torch.float64),
               self._empty(max(n, 1), torch.int32), self._empty(max(n, 1), torch.int32))
        if n == 0:
            coo[0].fill_(-1)
        check(lib.xmap_sim2_unpack_pairs(_stream(self.dev), i64(n), vp(rec.contiguous()), *[vp(x) for x in coo]))
        return [x[:max(n, 1)] for x in coo]

    def partial_records(self, coo, n, n_owners):
        """raw half COO (tri_pairs(raw=True)) -> [n][4] int64 records grouped by the rank that owns the pair's lower item"""
        st = _stream(self.dev)
        coo_i, coo_j, coo_hi, coo_mutu, coo_nij, coo_lo = coo
        rec = self._empty((max(n, 1), 4), torch.int64)
        cnt = C.c_int64(0)
        check(lib.xmap_sim2_pack_partials(st, i64(int(coo_i.numel())), vp(coo_i), vp(coo_j), vp(coo_hi), vp(coo_lo), vp(coo_mutu),
                                          vp(coo_nij), vp(rec), C.byref(cnt)))
        assert int(cnt.value) == n, (int(cnt.value), n)
        return self.sort_records(rec[:n], n_owners)

    def sort_records(self, rec, n_owners=0):
        """stable sort of partial records: by owner of the lower item (n_owners > 0) or by pair key"""
        n = int(rec.shape[0])
        out = self._empty((max(n, 1), 4), torch.int64)
        if n:
            check(lib.xmap_sim2_sort_partials(_stream(self.dev), i64(n), vp(rec.contiguous()), vp(out), i32(self.R.n_items),
                                              i32(n_owners)))
        return out[:n]

    def merge_records(self, rec_sorted, method, cap):
        """all records of the pairs this rank owns (sorted by key, equal keys in rank order) -> (coo, rowcnt, kept, evaluated)"""
        st = _stream(self.dev)
        I = self.R.n_items
        m = abi.METHODS[method] if isinstance(method, str) else int(method)
        n = int(rec_sorted.shape[0])
        coo_i = torch.full((max(n, 1),), -1, dtype=torch.int32, device=self.dev)
        coo_j = self._empty(max(n, 1), torch.int32)
        coo_sim = self._empty(max(n, 1), torch.float64)
        coo_mutu = self._empty(max(n, 1), torch.int32)
        coo_nij = self._empty(max(n, 1), torch.int32)
        rowcnt = self._empty(max(I, 1), torch.int32)
        h = (C.c_int64 * 2)()
        check(lib.xmap_sim2_merge_partials(st, m, int(cap), i32(I), i64(n), vp(rec_sorted.contiguous()), vp(self.norms), vp(coo_i),
                                           vp(coo_j), vp(coo_sim), vp(coo_mutu), vp(coo_nij), vp(rowcnt), h))
        kept, evaluated = int(h[0]), int(h[1])
        k1 = max(kept, 1)
        return (coo_i[:k1], coo_j[:k1], coo_sim[:k1], coo_mutu[:k1], coo_nij[:k1]), rowcnt, kept, evaluated

    def plan(self, slot_target=640):
        """work decomposition of the pair kernel: units = (item, hash partition of its partner space)"""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        Pn = SimResult()
        Pn.slot_target = slot_target
        Pn.Q = self._zeros(max(I, 1), torch.int32)
        Pn.W = self._zeros(max(I, 1), torch.int64)
        Pn.unit_ptr = self._zeros(I + 1, torch.int64)
        n_units, contrib = C.c_int64(0), C.c_int64(0)
        with self.timed("plan"):
            abi.xcheck(abi.xlib().xmap_sim_plan(st, C.byref(R.c), i32(slot_target), vp(Pn.Q), vp(Pn.W), vp(Pn.unit_ptr),
                                    C.byref(n_units), C.byref(contrib)))
        Pn.nu, Pn.contrib = int(n_units.value), int(contrib.value)
        Pn.unit_item = self._empty(max(Pn.nu, 1), torch.int32)
        Pn.unit_q = self._empty(max(Pn.nu, 1), torch.int32)
        abi.xcheck(abi.xlib().xmap_sim_units(st, i32(I), vp(Pn.Q), vp(Pn.unit_ptr), vp(Pn.unit_item), vp(Pn.unit_q)))
        return Pn

    def item_weights(self, Pn):
        """per-item cost proxy of the pair kernel: raters x partitions (rater-steps of its units)"""
        R = self.R
        n = (R.item_ptr[1:] - R.item_ptr[:-1])
        return n * Pn.Q[:R.n_items].to(torch.int64)

    def item_sim(self, method, cap, slot_target=SLOT_TARGET, item_range=None, stats=None, plan=None, algo="tri"):
        """baseliner_calculate_sim_pipeline.  algo "tri" (default): each unordered pair once + mirror
        (stage_a2.hip); algo "rows": complete rows per unit (stage_a.hip), supports item_range -- a TEST formulation: its
        kernels live in libxmap_hip_xcheck.so (abi.xlib()), not in the product library."""
        if algo == "tri" and item_range is None and plan is None:
            return self.item_sim_tri(method, cap, slot_target)
        R = self.R
        st = _stream(self.dev)
        m = abi.METHODS[method] if isinstance(method, str) else int(method)
        if stats is None or stats[3] is None:
            with self.timed("stats"):
                stats = self.stats(packed=True)
        u_avg, u_norm, info, ua_item, ia_user = stats
        I = R.n_items
        while True:
            Pn = plan if plan is not None else self.plan(slot_target)
            Q, unit_ptr, unit_item, unit_q, nu, contrib = Pn.Q, Pn.unit_ptr, Pn.unit_item, Pn.unit_q, Pn.nu, Pn.contrib
            if item_range is None:
                lo, hi = 0, nu
            else:
                ends = unit_ptr[[int(item_range[0]), int(item_range[1])]].tolist()
                lo, hi = int(ends[0]), int(ends[1])
            unit_cnt = self._zeros(max(nu, 1), torch.int32)
            d_cnt = self._zeros(4, torch.int64)
            h_cnt = [0, 0, 0, 0]
            with self.timed("pair_count"):
                rc = abi.xlib().xmap_sim_count(st, C.byref(R.c), m, int(cap), vp(u_avg), vp(info), vp(ua_item),
                                        vp(ia_user), vp(Q), vp(unit_item), vp(unit_q), i64(lo), i64(hi),
                                        vp(unit_cnt), vp(d_cnt), None)
            abi.xcheck(rc)
            h_cnt = d_cnt.tolist()  # synchronises the stream
            if h_cnt[2]:
                if slot_target > 32:
                    slot_target //= 2
                    plan = None
                    continue
                raise abi.XmapError(abi.ERR_OVERFLOW, "pair-table overflow")
            break
        kept, evaluated = int(h_cnt[0]), int(h_cnt[1])
        unit_off = self._zeros(nu + 1, torch.int64)
        check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(unit_cnt), vp(unit_off), i64(nu), None))
        row_ptr = self._empty(I + 1, torch.int64)
        abi.xcheck(abi.xlib().xmap_sim_row_ptr(st, i32(I), vp(unit_ptr), vp(unit_off), vp(row_ptr)))
        col = self._empty(max(kept, 1), torch.int32)
        sim = self._empty(max(kept, 1), torch.float64)
        mutu = self._empty(max(kept, 1), torch.int32)
        nij = self._empty(max(kept, 1), torch.int32)
        with self.timed("pair_fill"):
            abi.xcheck(abi.xlib().xmap_sim_fill(st, C.byref(R.c), m, int(cap), vp(u_avg), vp(info), vp(ua_item), vp(ia_user),
                                    vp(Q), vp(unit_item), vp(unit_q), i64(lo), i64(hi), vp(unit_off),
                                    vp(col), vp(sim), vp(mutu), vp(nij)))
        S = SimResult()
        S.method, S.cap, S.n_items = m, int(cap), I
        S.u_avg, S.u_norm, S.info = u_avg, u_norm, info
        S.row_ptr, S.col, S.sim, S.mutu, S.nij = row_ptr, col[:kept], sim[:kept], mutu[:kept], nij[:kept]
        S.n_kept, S.n_eval, S.n_contrib, S.n_units = kept, evaluated, int(contrib), hi - lo
        S.plan = Pn
        S.slot_target = slot_target
        S.c = abi.Sim(I, row_ptr.data_ptr(), col.data_ptr(), sim.data_ptr(), mutu.data_ptr(), nij.data_ptr(),
                      info.data_ptr(), 0)
        S._keep = (col, sim, mutu, nij)
        return S

    # ---- stage A, second formulation (stage_a2.hip): each unordered pair once, mirrored into the CSR
    def tri_layout(self, stats, slot_target=SLOT_TARGET, ch_min=CH_MIN, dups=False):
        """weight-sorted private profiles, rater records, heavy set, work units (method independent).
        dups: a profile may hold an item more than once (AlterEgo rows)."""
        R = self.R
        st = _stream(self.dev)
        I, U, nnz = R.n_items, R.n_users, R.nnz
        info = stats[2]
        L = SimResult()
        L.hist = self._empty(U + 2, torch.int32)
        L.pre = self._empty(U + 3, torch.int64)
        L.ctl = self._empty(4, torch.int32)
        L.hid = self._empty(max(I, 1), torch.int32)
        L.hlist = self._zeros(1024, torch.int32)
        L.ub_key = self._empty(max(nnz, 1), torch.int64)
        L.ub = self._empty(max(nnz, 1), torch.int64)           # (item | flag, rating bits) pairs
        L.rc = self._empty(max(nnz, 1) * 2, torch.int64)       # 16-byte rater records
        L.Wp = self._empty(max(I, 1), torch.int64)
        L.dups = bool(dups)
        h_ctl = (C.c_int32 * 2)()
        with self.timed("tri_layout"):
            check(lib.xmap_sim2_layout(st, C.byref(R.c), vp(info), i32(ch_min), vp(L.hist), vp(L.pre), vp(L.ctl),
                                       vp(L.hid), vp(L.hlist), vp(L.ub_key), vp(L.ub), vp(L.rc), vp(L.Wp),
                                       i32(1 if dups else 0), h_ctl))
        L.CH, L.n_heavy = int(h_ctl[0]), int(h_ctl[1])
        L.slot_target = slot_target
        self._tri_plan(L, slot_target)
        both = torch.stack([L.Wp[:I].sum(), L.Wp[L.hlist[:L.n_heavy].long()].sum()]).tolist() if I else [0, 0]   # one sync
        L.half_contrib, L.heavy_half = int(both[0]), int(both[1])
        return L

    def _tri_plan(self, L, slot_target):
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        L.Q = self._out(max(I, 1), torch.int32, I > 0)
        L.C = self._out(max(I, 1), torch.int32, I > 0)
        L.small = self._out(max(I, 1), torch.uint8, I > 0)
        L.Qcat = self._empty(5 * max(I, 1), torch.int32)
        L.uq_ptr = self._out(5 * I + 1, torch.int64, I > 0)     # light units class-major: [table class rank][item]
        L.uc_ptr = self._out(I + 1, torch.int64, I > 0)
        h = (C.c_int64 * 8)()
        with self.timed("tri_plan"):
            check(lib.xmap_sim2_plan(st, C.byref(R.c), i32(slot_target), vp(L.rc), vp(L.pre), vp(L.hid),
                                     vp(L.ctl), vp(L.Q), vp(L.C), vp(L.small), vp(L.Wp), vp(L.Qcat), vp(L.uq_ptr),
                                     vp(L.uc_ptr), i32(1 if getattr(L, "dups", False) else 0), h))
            L.n_light, L.n_heavy_units = int(h[0]), int(h[1])
            L.cls_ptr = (C.c_int64 * 6)(*[int(h[2 + c]) for c in range(6)])
            L.uq_item = self._empty(max(L.n_light, 1), torch.int32)
            L.uq_q = self._empty(4 * max(L.n_light, 1), torch.int32)
            L.uc_item = self._empty(max(L.n_heavy_units, 1), torch.int32)
            L.uc_c = self._empty(max(L.n_heavy_units, 1), torch.int32)
            check(lib.xmap_sim2_units(st, i32(I), vp(R.item_ptr), vp(L.Qcat), vp(L.uq_ptr), vp(L.uq_item), vp(L.uq_q),
                                      vp(L.C), vp(L.uc_ptr), vp(L.uc_item), vp(L.uc_c)))
        L.slot_target = slot_target

    def _tri_plan3(self, L, slot_target):
        """_tri_plan with one synchronisation (xmap_sim3_plan): unit arrays sized from host-known bounds, the counts the
        launches need and the heavy set's {CH, |H|} in one copy"""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        L.Q = self._out(max(I, 1), torch.int32, I > 0)
        L.C = self._out(max(I, 1), torch.int32, I > 0)
        L.small = self._out(max(I, 1), torch.uint8, I > 0)
        L.Qcat = self._empty(5 * max(I, 1), torch.int32)
        L.uq_ptr = self._out(5 * I + 1, torch.int64, I > 0)
        L.uc_ptr = self._out(I + 1, torch.int64, I > 0)
        cap_light = L.half_contrib // max(int(slot_target), 1) + I + 1
        cap_heavy = R.nnz // max(int(L.ch_min), 1) + 1025
        L.uq_item = self._empty(cap_light, torch.int32)
        L.uq_q = self._empty(4 * cap_light, torch.int32)
        L.uc_item = self._empty(cap_heavy, torch.int32)
        L.uc_c = self._empty(cap_heavy, torch.int32)
        h = (C.c_int64 * 10)()
        with self.timed("tri_plan"):
            check(lib.xmap_sim3_plan(st, C.byref(R.c), i32(slot_target), vp(L.pre), vp(L.hid), vp(L.ctl), vp(L.Q), vp(L.C),
                                     vp(L.small), vp(L.Wp), vp(L.Qcat), vp(L.uq_ptr), vp(L.uc_ptr),
                                     i32(1 if getattr(L, "dups", False) else 0), vp(L.uq_item), vp(L.uq_q), vp(L.uc_item),
                                     vp(L.uc_c), i64(cap_light), i64(cap_heavy), h))
        L.n_light, L.n_heavy_units = int(h[0]), int(h[1])
        L.cls_ptr = (C.c_int64 * 6)(*[int(h[2 + c]) for c in range(6)])
        L.CH, L.n_heavy = int(h[8]), int(h[9])
        L.slot_target = slot_target
        L.replan = self._tri_plan3

    def heavy_half(self, L):
        """contributions of the heavy rows (reporting only: the bench's per-kernel byte counts)"""
        if getattr(L, "_heavy_half", None) is None:
            L._heavy_half = int(L.Wp[L.hlist[:L.n_heavy].long()].sum().item()) if L.n_heavy else 0
        return L._heavy_half

    def tri_pairs(self, method, cap, stats, L, unit_range=None, do_heavy=True, retry=True, rec=False, raw=False, split=False,
                  heavy_deal=None, marks=True, count_mir=False):
        """half COO of the kept pairs computed by the light units in unit_range (+ the heavy rows).
        rec: the RecommenderSim variant (nothing filtered, self pairs, a 6th COO column with the local sensitivity).
        split: count a row's own pairs (rowcnt) and the pairs lighter rows computed for it (a 5th return value) apart --
        what tri_mirror takes.  heavy_deal = (rank, world): of the heavy rows only those with item index % world == rank
        (item-sharded ranks deal them round-robin; chunk partials and merge of a row stay on one rank).  split: the mirrored
        counts are NOT taken (the returned array is cleared): tri_mirror / mir_counts count them from the COO.  marks=False:
        the unused COO entries are left unmarked (for a consumer that goes by the shard cursors: tri_mirror with shards).
        count_mir (with split, all units on this device): the mirrored counts are taken here, launched BEFORE the read-back of
        the kept-pair count so that the host's wait falls into their 0.2 ms; tri_mirror is then called with counted=True."""
        R = self.R
        st = _stream(self.dev)
        m = abi.METHODS[method] if isinstance(method, str) else int(method)
        u_avg, u_norm, info, _, _ = stats
        I = R.n_items
        coo_slack = 1.0
        while True:
            lo, hi = (0, L.n_light) if unit_range is None else (int(unit_range[0]), int(unit_range[1]))
            # per-shard capacity >= one full table (1024 kept pairs) + the expected share with slack
            cap_coo = (max(int(L.half_contrib * coo_slack), 1) // 4096 + 1100) * 4096
            coo_i = self._empty(cap_coo, torch.int32)
            coo_j = self._empty(cap_coo, torch.int32)
            coo_sim = self._empty(cap_coo, torch.float64)
            coo_mutu = self._empty(cap_coo, torch.int32)
            coo_nij = self._empty(cap_coo, torch.int32)
            coo_ls = self._empty(cap_coo, torch.float64) if (rec or raw) else None    # raw: the error column of the dot
            rowcnt = self._empty(max(I, 1), torch.int32)
            mircnt = self._empty(max(I, 1), torch.int32) if split else None
            nh = L.n_heavy_units if do_heavy else 0
            hp_hi = self._empty(max(nh, 1) * 1024, torch.float64)
            hp_lo = self._empty(max(nh, 1) * 1024, torch.float64)
            hp_cnt = self._empty(max(nh, 1) * 1024, torch.int32)
            hp_mut = self._empty(max(nh, 1) * 1024, torch.int32)
            d_cnt = self._zeros(6, torch.int64)       # [4], [5]: the shard sums (phases bit 64)
            d_shards = self._empty(2 * 4096, torch.int64)
            rowcnt_h = self._empty(64 * 1024, torch.int32)

            deal = 0 if heavy_deal is None else ((int(heavy_deal[1]) & 0xff) << 16) | ((int(heavy_deal[0]) & 0xff) << 8)

            def run(phases):
                phases |= deal | (0 if marks else 128)
                check(lib.xmap_sim2_pairs(
                    st, C.byref(R.c), m, int(cap), vp(u_avg), vp(self.norms), vp(L.rc), vp(L.ub), vp(L.Q), vp(L.small),
                    vp(L.uq_item),
                    vp(L.uq_q), L.cls_ptr, i64(lo), i64(hi), vp(L.hid), vp(L.hlist), vp(L.ctl), vp(L.C), vp(L.uc_ptr),
                    vp(L.uc_item), vp(L.uc_c), i32(nh), i32(L.n_heavy), phases,
                    vp(hp_hi), vp(hp_lo), vp(hp_cnt), vp(hp_mut), i64(cap_coo), vp(coo_i), vp(coo_j), vp(coo_sim),
                    vp(coo_mutu), vp(coo_nij), vp(coo_ls), vp(rowcnt), vp(rowcnt_h), vp(d_shards), vp(d_cnt), vp(mircnt)))
            if os.environ.get("XMAP_SPLIT_PHASES") == "1":        # one timer per phase (analysis)
                with self.timed("pair_heavy"):
                    run(8 | (1 if do_heavy else 0))
                with self.timed("pair_tri"):
                    run(2 | (32 if raw else 0))
                with self.timed("heavy_merge"):
                    run((4 if do_heavy else 0) | 16)
            else:       # the heavy rows (chunk partials + merge) on a side stream next to the class launches of the light rows
                with self.timed("pair_tri"):
                    run(8 | 2 | 16 | 64 | (5 if do_heavy else 0) | (32 if raw else 0))
            if count_mir and split and not raw:
                with self.timed("mir_count"):
                    scratch = self._empty(cap_coo, torch.int32)
                    check(lib.xmap_sim3_mircount(st, i32(I), i64(cap_coo), vp(coo_i), vp(coo_j), vp(d_shards), i64(cap_coo),
                                                 i32(1 if rec else 0), vp(scratch), vp(mircnt)))
            h = d_cnt.tolist()          # the one synchronisation of the pair kernels: flags + kept / evaluated pairs
            if h[2]:
                if L.slot_target <= 32:
                    raise abi.XmapError(abi.ERR_OVERFLOW, "pair-table overflow")
                if not retry:      # sharded callers re-plan collectively
                    return (None, rowcnt, 0, 0) + ((None, None) if split else ()) + (1,)
                getattr(L, "replan", self._tri_plan)(L, L.slot_target // 2)
                continue
            if h[3]:            # a COO shard overflowed: more slack
                if coo_slack > 64:
                    raise abi.XmapError(abi.ERR_CAPACITY, "half-COO overflow")
                coo_slack *= 2
                continue
            break
        if os.environ.get("XMAP_SPLIT_PHASES") == "1":      # (the split run has no shard-sum phase)
            h[4:6] = d_shards.view(2, 4096).sum(dim=1).tolist()
        n, n_unordered = int(h[4]), int(h[5])
        coo = (coo_i, coo_j, coo_sim, coo_mutu, coo_nij) + ((coo_ls,) if (rec or raw) else ())
        out = (coo, rowcnt, n, n_unordered)
        if split:
            return out + (mircnt, d_shards) + (() if retry else (0,))
        return out if retry else out + (0,)

    def tri_scatter(self, coo, rowcnt, info, n=None, L=None):
        """mirror a (complete) half COO (n valid entries; unused ones have coo_i = -1) into the CSR"""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        coo_i, coo_j, coo_sim, coo_mutu, coo_nij = [x.contiguous() for x in coo[:5]]
        coo_ls = coo[5].contiguous() if len(coo) > 5 else None
        n_scan = int(coo_i.numel())
        if n is None:
            n = n_scan
        row_ptr = self._out(I + 1, torch.int64, I > 0)
        tot = C.c_int64(0)
        check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(rowcnt), vp(row_ptr), i64(I), C.byref(tot) if coo_ls is not None else None))
        kept = 2 * n if coo_ls is None else int(tot.value)      # a row paired with itself is one entry
        ls = self._empty(max(kept, 1), torch.float64) if coo_ls is not None else None
        col = self._empty(max(kept, 1), torch.int32)
        sim = self._empty(max(kept, 1), torch.float64)
        mutu = self._empty(max(kept, 1), torch.int32)
        nij = self._empty(max(kept, 1), torch.int32)
        fill = self._empty(max(I, 1), torch.int32)
        with self.timed("scatter"):
          if n:
            check(lib.xmap_sim2_scatter(st, i32(I), i64(n_scan), vp(coo_i), vp(coo_j), vp(coo_sim), vp(coo_mutu),
                                        vp(coo_nij), vp(coo_ls), vp(row_ptr), vp(fill), vp(L.hid), vp(L.hlist), vp(col),
                                        vp(sim), vp(mutu), vp(nij), vp(ls)))
        S = self.sim_from_device(row_ptr, col[:kept], sim[:kept], mutu[:kept], nij[:kept], info)
        S.ls = ls[:kept] if ls is not None else None
        return S

    def rec_sim(self, cap, slot_target=SLOT_TARGET):
        """RecommenderSim.calculate_sim (reference core/recommenderSim.py:65-133,188-195; both method names take the
        cosine branch) over this engine's ratings, which are AlterEgo rows: weighted cosine and leave-one-out local
        sensitivity of every directed item pair with a co-rater, CSR by first item (col, sim, nij, ls).  The same
        pair machinery as stage A: exact (double-double) sums with a zero user average, no heavy set."""
        R = self.R
        with self.timed("rec_stats"):
            stats, L = self.layout3(slot_target, ch_min=max(64, R.n_users + 2), wide=True)      # no heavy set
        info = stats[2]
        if os.environ.get("XMAP_A_V2") == "1":        # round-2 mirror (cursor atomics), kept as a cross-check
            coo, rowcnt, n, n_unordered = self.tri_pairs("adjust_cosine", cap, stats, L, do_heavy=False, rec=True)
            S = self.tri_scatter(coo, rowcnt, info, n, L)
        else:
            coo, own, n, n_unordered, mir, shards = self.tri_pairs("adjust_cosine", cap, stats, L, do_heavy=False, rec=True, split=True, marks=False, count_mir=True)
            S = self.tri_mirror(coo, own, mir, info, n, shards, counted=True)
        S.cap, S.n_unordered, S.layout = int(cap), n_unordered, L
        S.norm = self.norms[R.n_items:2 * R.n_items]
        return S

    def rec_select(self, S, keep):
        """RecommenderPrivacy.nonprivate_neighbor_selection on a rec_sim result: per item the `keep` neighbours by
        (|sim| desc, index asc).  Returns (cnt [I], col [I][keep], sim [I][keep], ls [I][keep]) on the device."""
        st = _stream(self.dev)
        I = self.R.n_items
        cnt = self._zeros(max(I, 1), torch.int32)
        col = self._empty((max(I, 1), keep), torch.int32)
        sim = self._empty((max(I, 1), keep), torch.float64)
        ls = self._empty((max(I, 1), keep), torch.float64)
        with self.timed("rec_select"):
            check(lib.xmap_rec_select(st, i32(I), vp(S.row_ptr), vp(S.col), vp(S.sim), vp(S.ls), i32(keep), vp(cnt),
                                      vp(col), vp(sim), vp(ls)))
        return cnt[:I], col[:I], sim[:I], ls[:I]

    def layout3(self, slot_target=SLOT_TARGET, ch_min=CH_MIN, wide=False, item_range=None):
        """Round-3 layout of the "tri" formulation, one transposition per pass (xmap_sim3_layout): item counts, user and
        item info, weight-sorted profiles, rater records through the tile sort, heavy set, work units.  Returns (stats, L)
        like stats() + tri_layout().  wide: fp64 ratings (R.user_rating64) with zero user averages -- the RecommenderSim
        variant.  item_range (item-sharded ranks): the item statistics of that share of the items only; the caller then
        completes stats[2] (info) and self.norms on every rank -- the all-gather of per-item norms -- and calls L.finish(),
        which sets the mutuality flags from them and plans the work units."""
        R = self.R
        st = _stream(self.dev)
        I, U, nnz = R.n_items, R.n_users, R.nnz
        n1, i1 = max(nnz, 1), max(I, 1)
        rw = 3 if wide else 2                                    # 64-bit words per sort record
        L = SimResult()
        cnt = self._empty(i1, torch.int32)
        u_avg = self._zeros(max(U, 1), torch.float64) if wide else self._empty(max(U, 1), torch.float64)
        u_norm = None if wide else self._empty(max(U, 1), torch.float64)
        L.hist = self._empty(U + 2, torch.int32)
        L.pre = self._empty(U + 3, torch.int64)
        L.ctl = self._empty(4, torch.int32)
        L.hid = self._empty(i1, torch.int32)
        L.hlist = self._zeros(1024, torch.int32)
        L.ub_key = self._empty(n1, torch.int64)
        L.ub = self._empty(n1 * (2 if wide else 1), torch.int64)
        srec = self._empty(n1 * rw, torch.int64)
        buf_a = self._empty(n1 * rw, torch.int64)
        buf_b = self._empty(n1 * rw, torch.int64)
        L.rc = self._empty(n1 * 2, torch.int64)
        L.Wp = self._empty(i1, torch.int64)
        L.dups = bool(wide)
        L.wide = bool(wide)
        info = self._out((i1, 4), torch.float64, I > 0)
        self.norms = self._out(2 * i1, torch.float64, I > 0)
        r64 = None
        if wide:
            r64 = R.user_rating64 if R.user_rating64 is not None else R.user_rating.double()
        def call(phases, lo, hi):
            check(lib.xmap_sim3_layout(st, C.byref(R.c), vp(R.item_ptr), vp(r64), i32(ch_min), i32(phases), i32(lo), i32(hi), vp(cnt),
                                       vp(u_avg), vp(u_norm), vp(L.hist), vp(L.pre), vp(L.ctl), vp(L.hid), vp(L.hlist),
                                       vp(L.ub_key), vp(L.ub), vp(srec), vp(buf_a), vp(buf_b), vp(L.rc), vp(L.Wp), vp(info),
                                       vp(self.norms), None))
        R.csc_ready = False             # item_ptr is current; item_user / item_rating are not built on this path
        L.ch_min = int(ch_min)
        L.half_contrib = R.half_contrib          # = sum of W+ over the items (the host knows it from the profile lengths)
        if item_range is None:
            with self.timed("layout3"):
                call(1 | 2 | 4, 0, I)
            self._tri_plan3(L, slot_target)
        else:
            with self.timed("layout3"):
                call(1 | 2, int(item_range[0]), int(item_range[1]))

            def finish():
                with self.timed("layout3_flags"):
                    call(4 | 8, 0, I)
                self._tri_plan3(L, slot_target)
            L.finish = finish
        return (u_avg, u_norm, info, None, None), L

    def mir_counts(self, coo, n, mir, shards=None, skip_self=False):
        """mir[j] = entries of the half COO (n valid ones) whose partner is j (xmap_sim3_mircount)"""
        I = self.R.n_items
        scratch = self._empty(max(int(n), 1), torch.int32)
        check(lib.xmap_sim3_mircount(_stream(self.dev), i32(I), i64(int(coo[0].numel())), vp(coo[0].contiguous()), vp(coo[1].contiguous()),
                                     vp(shards), i64(n), i32(1 if skip_self else 0), vp(scratch), vp(mir)))
        return mir

    def tri_mirror(self, coo, own, mir, info, n, shards=None, rows=None, counted=False):
        """round-3 mirror (xmap_sim3_mirror): a complete half COO with n valid entries, own[i] = pairs row i computed,
        mir[j] = pairs computed in lighter rows -> CSR, row = [own | mirrored].  shards: the cursors tri_pairs left (the
        COO is cut into 4096 shards filled from their start); None: the COO is one range of n records.  rows = (lo, hi):
        only these rows are built (the others stay empty) -- an item-sharded rank's share of the matrix, which is all its
        knn and reverse-list shares read.  mir is filled here from the COO unless counted=True (mir_counts ran already)."""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        coo_i, coo_j, coo_sim, coo_mutu, coo_nij = [x.contiguous() for x in coo[:5]]
        coo_ls = coo[5].contiguous() if len(coo) > 5 else None       # RecommenderSim: local sensitivities, self pairs
        cap = int(coo_i.numel())
        kept = 2 * int(n)                                            # (an upper bound with self pairs: one entry each)
        lo, hi = (0, I) if rows is None else (int(rows[0]), int(rows[1]))
        if rows is not None and I > 0:
            own = own.clone()
            own[:lo].zero_(); own[hi:].zero_()
            if counted:
                mir = mir.clone()
                mir[:lo].zero_(); mir[hi:].zero_()
                kept = int((own[lo:hi].long().sum() + mir[lo:hi].long().sum()).item())
        rw = 4 if coo_ls is not None else 3
        row_ptr = self._out(I + 1, torch.int64, I > 0)
        mptr = self._empty(I + 1, torch.int64)
        tot = self._empty(max(I, 1), torch.int32)
        fill = self._empty(max(I, 1), torch.int32)
        buf_a = self._empty(max(int(n), 1) * rw, torch.int64)
        buf_b = self._empty(max(int(n), 1) * rw, torch.int64)
        col = self._empty(max(kept, 1), torch.int32)
        sim = self._empty(max(kept, 1), torch.float64)
        mutu = self._empty(max(kept, 1), torch.int32)
        nij = self._empty(max(kept, 1), torch.int32)
        ls = self._empty(max(kept, 1), torch.float64) if coo_ls is not None else None
        with self.timed("scatter"):
            if not counted:
                check(lib.xmap_sim3_mircount(st, i32(I), i64(cap), vp(coo_i), vp(coo_j), vp(shards), i64(n), i32(1 if coo_ls is not None else 0),
                                             vp(buf_a), vp(mir)))
                if rows is not None and I > 0:
                    mir[:lo].zero_(); mir[hi:].zero_()
                    kept = int((own[lo:hi].long().sum() + mir[lo:hi].long().sum()).item())
            check(lib.xmap_sim3_mirror(st, i32(I), i64(cap), vp(coo_i), vp(coo_j), vp(coo_sim), vp(coo_mutu), vp(coo_nij), vp(shards), i64(n),
                                       vp(own), vp(mir), vp(tot), vp(row_ptr), vp(mptr), vp(fill), vp(buf_a), vp(buf_b), vp(col),
                                       vp(sim), vp(mutu), vp(nij), vp(coo_ls), vp(ls), i32(lo), i32(hi)))
        if coo_ls is not None:
            kept = int(row_ptr[I].item()) if I > 0 else 0
        S = self.sim_from_device(row_ptr, col[:kept], sim[:kept], mutu[:kept], nij[:kept], info)
        S.ls = ls[:kept] if ls is not None else None
        return S

    def item_sim_tri(self, method, cap, slot_target=SLOT_TARGET, ch_min=CH_MIN):
        """baseliner_calculate_sim_pipeline, second formulation (all rows, one GPU).  XMAP_A_V2=1: the round-2 sequence
        (CSC build, CSC-driven rater records, cursor-atomic mirror) -- kept as a cross-check of the round-3 one."""
        if os.environ.get("XMAP_A_V2") == "1":
            with self.timed("stats"):
                stats = self.stats()
            L = self.tri_layout(stats, slot_target, ch_min)
            coo, rowcnt, n, n_unordered = self.tri_pairs(method, cap, stats, L)
            S = self.tri_scatter(coo, rowcnt, stats[2], n, L)
        else:
            stats, L = self.layout3(slot_target, ch_min)
            coo, rowcnt, n, n_unordered, mir, shards = self.tri_pairs(method, cap, stats, L, split=True, marks=False, count_mir=True)
            S = self.tri_mirror(coo, rowcnt, mir, stats[2], n, shards, counted=True)
        S.method = abi.METHODS[method] if isinstance(method, str) else int(method)
        S.cap = int(cap)
        S.u_avg, S.u_norm = stats[0], stats[1]
        S.n_eval, S.n_contrib = 2 * n_unordered, 2 * L.half_contrib
        S.n_units, S.layout, S.slot_target = L.n_light + L.n_heavy_units, L, L.slot_target
        return S

    def sim_from_device(self, row_ptr, col, sim, mutu, nij, info):
        """Wrap device tensors (e.g. rows gathered from several ranks) as a SimResult."""
        S = SimResult()
        S.n_items = self.R.n_items
        n = int(col.numel())
        if n == 0:
            col = self._zeros(1, torch.int32); sim = self._zeros(1, torch.float64)
            mutu = self._zeros(1, torch.int32); nij = self._zeros(1, torch.int32)
        S.row_ptr, S.info = row_ptr.contiguous(), info
        col, sim, mutu, nij = col.contiguous(), sim.contiguous(), mutu.contiguous(), nij.contiguous()
        S.col, S.sim, S.mutu, S.nij = col[:n], sim[:n], mutu[:n], nij[:n]
        S.n_kept = n
        S.c = abi.Sim(S.n_items, S.row_ptr.data_ptr(), col.data_ptr(), sim.data_ptr(), mutu.data_ptr(),
                      nij.data_ptr(), info.data_ptr())
        S._keep = (col, sim, mutu, nij)
        return S

    def sim_from_host(self, row_ptr, col, sim, mutu, nij, info, frac=None):
        """Wrap a host-side stage-A result (e.g. a canonically re-fed RDD) as a device SimResult.
        With `frac`, frac_mutu is taken from the records instead of being derived from (info, nij)."""
        d = self.dev
        n = int(row_ptr[-1])

        def up(a, dt):
            return torch.from_numpy(np.ascontiguousarray(a if n else np.zeros(1), dt)).to(d)
        row_ptr_d = torch.from_numpy(np.ascontiguousarray(row_ptr, np.int64)).to(d)
        info_d = torch.from_numpy(np.ascontiguousarray(info, np.float64)).to(d)
        S = self.sim_from_device(row_ptr_d, up(col, np.int32)[:n], up(sim, np.float64)[:n], up(mutu, np.int32)[:n],
                                 up(nij if nij is not None else np.zeros(max(n, 1)), np.int32)[:n], info_d)
        if frac is not None:
            S.frac = up(frac, np.float64)
            S.c.frac = S.frac.data_ptr()
        return S

    # ------------------------------------------------------------------ stage B
    def knn(self, S, top_k, bb=None, rows=None):
        """B1-B4: bridge flags (computed, or given by the caller) + classified top-k lists.  rows = (lo, hi): the lists of
        that share of the items only (item-sharded ranks all-gather the tables)."""
        R = self.R
        st = _stream(self.dev)
        I, k = R.n_items, int(top_k)
        E = ExtResult()
        E.k = k
        if bb is not None:
            E.bb = bb.to(torch.uint8).contiguous()
        else:
            E.bb = self._zeros(max(I, 1), torch.uint8)
            check(lib.xmap_bridge_flags(st, C.byref(S.c), vp(R.prefix_cls), vp(E.bb)))
        alloc = self._empty if I > 0 else self._zeros      # xmap_knn_classify writes every entry of the rows it is given
        E.cls = alloc(max(I, 1), torch.uint8)                # (item-sharded ranks all-gather the other rows: ext_gather)
        E.kcnt = alloc((max(I, 1), 2), torch.int32)
        E.kcol = alloc((max(I, 1), 2, k), torch.int32)
        E.kval = alloc((max(I, 1), 2, k, 3), torch.float64)
        with self.timed("knn_classify"):
            lo, hi = (0, I) if rows is None else (int(rows[0]), int(rows[1]))
            check(lib.xmap_knn_classify(st, C.byref(S.c), k, vp(E.bb), vp(R.suffix_cls), vp(R.contains_mask),
                                        vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval), i32(lo), i32(hi)))
        return E

    def _reverse(self, S, E, mode, attach_ptr):
        """one reverse adjacency (mode 0 attach, 1 src, 2 rnn), all rows on this device"""
        st8 = self.reverse_count(S, E, mode, attach_ptr, None)
        self.reverse_fill(st8)
        return st8.out

    def reverse_count(self, S, E, mode, attach_ptr, rows):
        """count pass of one reverse adjacency over the rows [lo, hi) (None: all).  The list of a row is built from that row
        of the similarity matrix alone (the matrix is symmetric: "b lists a" is tested on a's own entry for b), so a rank's
        share of the rows is a contiguous share of the lists -- item-sharded ranks all-gather the pieces (reverse_gather_*)."""
        R = self.R
        I = R.n_items
        st8 = ExtResult()
        st8.S, st8.E, st8.mode = S, E, mode
        st8.rows = (0, I) if rows is None else (int(rows[0]), int(rows[1]))
        st8.rcnt = self._zeros(max(I, 1), torch.int32)
        # one byte per entry of these rows: what the count pass finds, for the fill pass (which then gathers nothing)
        st8.eflag = self._empty(max(int(S.col.numel()), 1), torch.uint8)       # (indexed from the first entry of row lo)
        st8.args = (C.byref(S.c), mode, E.k, vp(E.bb), vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval),
                    vp(R.suffix_cls), vp(R.contains_mask), vp(R.flags), vp(attach_ptr), vp(getattr(E, "thr", None)),
                    vp(getattr(E, "long_rows", None)), vp(st8.eflag))
        st8.keep = attach_ptr
        check(lib.xmap_reverse_count(_stream(self.dev), *st8.args, vp(st8.rcnt), i32(st8.rows[0]), i32(st8.rows[1])))
        return st8

    def reverse_count_pair(self, S, E, rows):
        """the count passes of the attach (mode 0) and rnn (mode 2) lists as ONE pass over the rows (both ask the non-bridge
        neighbours of an entry, about their two lists): two handles like reverse_count's, sharing the per-entry bytes"""
        R = self.R
        I = R.n_items
        lo, hi = (0, I) if rows is None else (int(rows[0]), int(rows[1]))
        eflag = self._empty(max(int(S.col.numel()), 1), torch.uint8)
        sts = []
        for mode in (0, 2):
            st8 = ExtResult()
            st8.S, st8.E, st8.mode, st8.rows = S, E, mode, (lo, hi)
            st8.rcnt = self._zeros(max(I, 1), torch.int32)
            st8.eflag = eflag
            st8.args = (C.byref(S.c), mode, E.k, vp(E.bb), vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval),
                        vp(R.suffix_cls), vp(R.contains_mask), vp(R.flags), vp(None), vp(E.thr),
                        vp(getattr(E, "long_rows", None)), vp(eflag))
            st8.keep = None
            sts.append(st8)
        check(lib.xmap_reverse_count_att_rnn(_stream(self.dev), C.byref(S.c), E.k, vp(E.bb), vp(E.cls), vp(E.kcnt), vp(E.kcol),
                                             vp(E.kval), vp(R.suffix_cls), vp(R.contains_mask), vp(R.flags), vp(E.thr),
                                             vp(getattr(E, "long_rows", None)), vp(eflag), vp(sts[0].rcnt), vp(sts[1].rcnt),
                                             i32(lo), i32(hi)))
        return sts

    def reverse_gather_counts(self, sts, comm):
        """collective: every rank's counts of its rows -> the counts of all rows (of several lists in one exchange)"""
        I = self.R.n_items
        lo, hi = sts[0].rows
        for st8, c in zip(sts, comm.all_gather_multi([st8.rcnt[lo:hi] for st8 in sts])):
            st8.rcnt[:I] = c

    def reverse_fill(self, st8):
        """offsets of all rows (scan of the complete counts), then the entries of this device's rows"""
        I = self.R.n_items
        st = _stream(self.dev)
        rptr = self._zeros(I + 1, torch.int64)
        tot = C.c_int64(0)
        check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(st8.rcnt), vp(rptr), i64(I), C.byref(tot)))
        n = int(tot.value)
        ridx = self._empty(max(n, 1), torch.int32)
        rval = self._empty((max(n, 1), 3), torch.float64)
        rflag = self._zeros(max(n, 1), torch.uint8)
        check(lib.xmap_reverse_fill(st, *st8.args, vp(rptr), vp(ridx), vp(rval), vp(rflag), i32(st8.rows[0]), i32(st8.rows[1])))
        st8.out = (rptr, ridx, rval, rflag, n)

    def reverse_gather(self, sts, comm):
        """collective: the ranks' pieces of the lists, in rank (= row) order -- ONE exchange for all the lists given"""
        send = []
        for st8 in sts:
            rptr, ridx, rval, rflag, n = st8.out
            lo, hi = st8.rows
            ends = rptr[[lo, hi]].tolist()
            a, b = int(ends[0]), int(ends[1])
            send += [ridx[a:b], rval[a:b].reshape(-1), rflag[a:b]]
        got = comm.all_gather_multi(send)
        for j, st8 in enumerate(sts):
            rptr, ridx, rval, rflag, n = st8.out
            if n:
                ridx[:n] = got[3 * j]
                rval[:n] = got[3 * j + 1].view(n, 3)
                rflag[:n] = got[3 * j + 2]
        return [st8.out for st8 in sts]

    def path_units(self, E, start_range=None, chunk=None, row_budget=48 << 30, start_split=None):
        """Work units of the path enumeration from the exact per-start path counts: starts with more than
        `chunk` paths are split into G round-robin chunks with dedicated accumulator rows (merged on the
        device afterwards); units are ordered heaviest first."""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        tmp = self._zeros(4 * max(I, 1), torch.int64)
        P = self._zeros(max(I, 1), torch.int64)
        with self.timed("path_weights"):
            check(lib.xmap_path_weights(st, i32(I), E.k, vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(R.flags),
                                        vp(E.att[0]), vp(E.att[1]), vp(E.src[0]), vp(E.src[1]), vp(E.src[3]),
                                        vp(E.rnn[0]), vp(E.rnn[1]), vp(tmp), vp(P)))
        # planning in the library (xmap_path_plan: chunk counts, heaviest-first order by its own radix sort, unit arrays)
        if start_split is not None:   # (rank, world): contiguous start ranges of equal path counts
            from .sharded import balanced_ranges
            start_range = balanced_ranges(P[:I].cpu().numpy(), start_split[1])[start_split[0]]
        lo, hi = (0, I) if start_range is None else (int(start_range[0]), int(start_range[1]))
        row_bytes = 36 * max(I, 1)
        row_budget = int(float(os.environ.get("XMAP_ROW_BUDGET_GB", row_budget / (1 << 30))) * (1 << 30))
        row_budget = min(row_budget, int(0.15 * self.hbm_available("qhacc", "hacc")))      # (the rows of the split heavy starts)
        max_rows = max(row_budget // row_bytes, 2)
        cap_units = max(I, 1) + max_rows
        U = ExtResult()
        U.unit_start = self._empty(cap_units, torch.int32)
        U.unit_c = self._empty(cap_units, torch.int32)
        U.unit_G = self._empty(cap_units, torch.int32)
        U.unit_row = self._empty(cap_units, torch.int32)
        U.heavy_unit0 = self._empty(max(I, 1), torch.int32)
        h = (C.c_int64 * 5)()
        with self.timed("path_plan"):
            check(lib.xmap_path_plan(st, i32(I), vp(P), i32(lo), i32(hi), i64(chunk or 0),
                                     i64(0 if chunk else int(os.environ.get("XMAP_CHUNK_DIV", "8192"))), i64(max_rows),
                                     i64(cap_units), vp(U.unit_start), vp(U.unit_c), vp(U.unit_G), vp(U.unit_row),
                                     vp(U.heavy_unit0), h))
        U.n_units, U.n_heavy, U.n_rows, U.total, U.chunk = int(h[0]), int(h[1]), int(h[2]), int(h[3]), int(h[4])
        U.P = P          # exact path count per start (B5d / B5e)
        # all starts, not only this call's range: what the refusal of extend_tables goes by, so that the ranks of a sharded
        # step decide alike and the message carries the whole problem's count
        U.total_all = int(P[:I].sum().item()) if (I and (start_range is not None or start_split is not None)) else U.total
        U.unit_nt = self._zeros(max(U.n_units, 1), torch.int32)
        return U

    def end_universe(self, E):
        """items that can end a path, ranked by item index: E.urank [I] (-1: not an end), E.uitem [n_ends]"""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        T = self._ext_tables(E, None)
        mark = self._empty(max(I, 1), torch.int32)
        rank = self._empty(I + 1, torch.int64)
        E.urank = self._empty(max(I, 1), torch.int32)
        E.uitem = self._empty(max(I, 1), torch.int32)
        n = C.c_int64(0)
        check(lib.xmap_end_universe(st, C.byref(T), vp(mark), vp(rank), vp(E.urank), vp(E.uitem), C.byref(n)))
        E.n_ends = int(n.value)
        M = getattr(E, "mid", None)
        if os.environ.get("XMAP_URANK_HOME", "1") == "1" and E.n_ends and M is not None:
            # row order: the ends of a column side by side (a column's row update then touches few lines)
            check(lib.xmap_end_order(st, i32(I), E.k, i32(M.n_nb), vp(M.nb_list), vp(E.kcnt), vp(E.kcol), i32(E.n_ends),
                                     vp(E.urank), vp(E.uitem)))

    def _ext_tables(self, E, M):
        R = self.R
        p = lambda t: (t.data_ptr() if t is not None else 0)
        g = lambda o, n: getattr(o, n, None) if o is not None else None
        return abi.ExtTables(R.n_items, E.k, p(E.cls), p(E.kcnt), p(E.kcol), p(E.kval), p(R.flags),
                             p(E.att[0]), p(E.att[1]), p(E.att[2]), p(E.src[0]), p(E.src[1]), p(E.src[2]), p(E.src[3]),
                             p(E.rnn[0]), p(E.rnn[1]), p(E.rnn[2]),
                             (M.n_nb if M is not None else 0), p(g(M, "nb_id")), p(g(M, "nb_list")), p(g(M, "midX")),
                             p(g(M, "dir")), p(g(M, "dir_ptr")),
                             getattr(E, "n_ends", 0), p(getattr(E, "urank", None)), p(getattr(E, "uitem", None)))

    def mid_lists(self, E, table_budget=24 << 30, rows=None):
        """middle lists of all joint paths, one tile of records per (x', x) (stage_b.hip, second formulation).
        rows (default): built row-wise, one block per x' with the row's tile sizes in LDS (abi.MID_ROWS_SPAN columns at a
        time); rows=False / XMAP_MID_TABLE=1: through the dense n_nb x n_nb tile table (cross-check; None when it does not
        fit the budget -- callers fall back to the per-path enumeration)."""
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        nb_list = self._empty(max(I, 1), torch.int32)
        nb_id = self._empty(max(I, 1), torch.int32)
        nn = C.c_int64(0)
        check(lib.xmap_nb_index(st, i32(I), vp(E.cls), vp(nb_list), vp(nb_id), C.byref(nn)))
        n_nb = int(nn.value)
        nb_list = nb_list[:n_nb]
        if rows is None:
            rows = os.environ.get("XMAP_MID_TABLE") != "1"
        if n_nb == 0 or (not rows and n_nb * n_nb * 12 > table_budget):
            return None
        M = ExtResult()
        M.n_nb, M.nb_list = n_nb, nb_list
        M.nb_id = nb_id
        common = (i32(I), E.k, vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval), vp(R.flags), vp(E.att[0]), vp(E.att[1]),
                  vp(E.att[2]), vp(E.src[0]), vp(E.src[1]), vp(E.src[2]), vp(E.src[3]), i32(n_nb), vp(M.nb_list),
                  vp(M.nb_id))
        if rows:
            with self.timed("mid_build"):
                M.ng = self._empty(n_nb, torch.int32)
                nrec = self._empty(n_nb, torch.int64)
                check(lib.xmap_mid_rows_count(st, *common, vp(M.ng), vp(nrec)))
                M.dir_ptr = self._zeros(n_nb + 1, torch.int64)
                rec_ptr = self._zeros(n_nb + 1, torch.int64)
                tx, tg = C.c_int64(0), C.c_int64(0)
                check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(M.ng), vp(M.dir_ptr), i64(n_nb), C.byref(tg)))
                check(lib.xmap_exclusive_scan_i64(st, vp(nrec), vp(rec_ptr), i64(n_nb), C.byref(tx)))
                M.n_records, M.n_tiles = int(tx.value), int(tg.value)
                budget = float(os.environ.get("XMAP_MID_BUDGET_GB", "100")) * 1e9
                if 64.0 * M.n_records + 24.0 * M.n_tiles > budget:
                    return None         # the lists do not fit: the caller enumerates path by path
                M.dir = self._empty(max(M.n_tiles, 1) * 3, torch.int64)
                M.midX = self._empty(max(M.n_records, 1) * 8, torch.float64)
                check(lib.xmap_mid_rows_place(st, *common, vp(M.dir_ptr), vp(rec_ptr), vp(M.dir), vp(M.midX)))
            return M
        with self.timed("mid_build"):
            tile_cnt = self._empty(n_nb * n_nb, torch.int32)
            M.ng = self._zeros(n_nb, torch.int32)
            abi.xcheck(abi.xlib().xmap_mid_tally(st, *common, vp(tile_cnt), vp(M.ng)))
            tile_off = self._empty(n_nb * n_nb + 1, torch.int64)
            M.dir_ptr = self._zeros(n_nb + 1, torch.int64)
            tx, tg = C.c_int64(0), C.c_int64(0)
            check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(tile_cnt), vp(tile_off), i64(n_nb * n_nb), C.byref(tx)))
            check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(M.ng), vp(M.dir_ptr), i64(n_nb), C.byref(tg)))
            M.n_records, M.n_tiles = int(tx.value), int(tg.value)
            M.dir = self._empty(max(M.n_tiles, 1) * 3, torch.int64)
            M.midX = self._empty(max(M.n_records, 1) * 8, torch.float64)
            abi.xcheck(abi.xlib().xmap_mid_place(st, *common, vp(tile_cnt), vp(tile_off), vp(M.dir_ptr), vp(M.dir), vp(M.midX)))
        return M

    def _extend_cols(self, E, U, M, full, xs_cap, start_range, n_slots):
        """xmap_extend_cols: one set of lanes and one row update per column, rows indexed by end rank"""
        R = self.R
        I = R.n_items
        st = _stream(self.dev)
        with self.timed("paths_prep"):
            self.end_universe(E)
        nU = max(E.n_ends, 1)
        # the accumulator rows (36 B x ends each: one per resident wave) take what the device can spare: 45 % of the HBM this
        # process could still get (free + what its caching allocator holds unused + the rows of an earlier pass), at most
        # XMAP_SLOT_BUDGET_GB -- fewer rows only mean fewer waves in flight (the S1 shape of the reference's report, 5.3e5
        # ends, peaked at 129 GB with a fixed 120 GB allowance)
        slot_budget = min(int(float(os.environ.get("XMAP_SLOT_BUDGET_GB", "120")) * (1 << 30)), int(0.45 * self.hbm_available("qacc")))
        if "XMAP_N_SLOTS" in os.environ:
            n_slots = int(os.environ["XMAP_N_SLOTS"])
        else:       # one row per wave the kernel keeps resident (4 per SIMD since round 4)
            ns = C.c_int32(0)
            check(lib.xmap_extend_cols_slots(C.byref(ns)))
            n_slots = min(n_slots, int(ns.value))
        n_slots = int(max(4, min(n_slots, slot_budget // (36 * nU), max(U.n_units, 4))))
        abl = int(os.environ.get("XMAP_ABL_ROW_ENTRIES", "0"))     # (ablation builds of k_paths4 with longer rows, -DQ_STORE)
        acc = self._zero_scratch("qacc", n_slots * max(nU, abl) * 4, torch.float64)
        touched = self._empty(n_slots * nU, torch.int32)
        hacc = self._zero_scratch("qhacc", max(U.n_rows, 1) * nU * 4, torch.float64) if U.n_rows else None
        htouched = self._empty(max(U.n_rows, 1) * nU, torch.int32) if U.n_rows else None
        E.n_cand = self._zeros(max(I, 1), torch.int32)
        E.top_end = torch.full((max(I, 1), abi.TOPC), -1, dtype=torch.int32, device=self.dev)
        E.top_val = self._zeros((max(I, 1), abi.TOPC), torch.float64)
        d_cnt = self._zeros(8, torch.int64)
        h_cnt = (C.c_int64 * 8)()
        T = self._ext_tables(E, M)
        Un = abi.PathUnits(U.n_units, U.unit_start.data_ptr(), U.unit_c.data_ptr(), U.unit_G.data_ptr(), U.unit_row.data_ptr(),
                           U.unit_nt.data_ptr(), U.n_heavy, U.heavy_unit0.data_ptr())
        Rw = abi.PathRows(n_slots, acc.data_ptr(), touched.data_ptr(), hacc.data_ptr() if hacc is not None else 0,
                          htouched.data_ptr() if htouched is not None else 0)
        cap = 0
        if full:
            cap = int(xs_cap) if xs_cap else 1 << 22
        fast = 0 if os.environ.get("XMAP_SLOW_DIV") == "1" else int(getattr(E, "fast_div", 0))
        while True:
            xs_off = self._zeros(max(I, 1), torch.int64) if cap else None
            xs_end = self._empty(max(cap, 1), torch.int32) if cap else None
            xs_val = self._empty(max(cap, 1), torch.float64) if cap else None
            O = abi.PathOut(E.n_cand.data_ptr(), E.top_end.data_ptr(), E.top_val.data_ptr(), cap,
                            xs_off.data_ptr() if cap else 0, xs_end.data_ptr() if cap else 0, xs_val.data_ptr() if cap else 0)
            try:
                with self.timed("paths"):
                    rc = lib.xmap_extend_cols(st, C.byref(T), C.byref(Un), C.byref(Rw), C.byref(O), fast, vp(d_cnt), h_cnt)
                if rc == abi.ERR_CAPACITY:      # the pass itself completed (rows are back to zero): lists did not fit
                    cap = int(h_cnt[0])
                    continue
                check(rc)
            except BaseException:
                self._drop_scratch("qacc", "qhacc")    # a failed pass may leave partial sums behind
                raise
            break
        E.n_out, E.n_paths, E.n_updates = int(h_cnt[0]), int(h_cnt[1]), int(h_cnt[4])
        E.row_info = dict(n_slots=n_slots, ends=nU, slot_rows_gb=n_slots * nU * 36 / 1e9, heavy_rows=int(U.n_rows),
                          heavy_rows_gb=int(U.n_rows) * nU * 36 / 1e9, slot_budget_gb=slot_budget / 1e9)
        E.xs_off, E.xs_end, E.xs_val = xs_off, xs_end, xs_val
        E.start_range = (0, I) if start_range is None else tuple(start_range)
        return E

    def ext_tables(self, S, top_k, comm=None):
        """B1-B5b: bridge flags, classified top-k lists and the three reverse adjacencies (attach / src / rnn).
        comm (xmap.engine.sharded.Comm of several ranks): every rank classifies a share of the rows -- contiguous ranges of
        equal entry counts -- and the tables are all-gathered (the reference broadcasts them, utils/assist.py:93-95).
        Three phases (a sharded caller puts its collective error check between them: ext_knn and ext_reverse may raise,
        ext_gather holds every collective and nothing else)."""
        E = self.ext_knn(S, top_k, comm)
        self.ext_gather(E, comm)
        return self.ext_reverse(S, E)

    def row_shares(self, tot, world):
        """cut points of `world` contiguous row shares of (about) equal entries; tot [I] = entries per row"""
        I = self.R.n_items
        cum = torch.cumsum(tot[:I].long(), 0)
        tgt = (cum[I - 1].double() * torch.arange(1, world, dtype=torch.float64, device=self.dev) / world).long()
        cuts = [0] + (torch.searchsorted(cum, tgt, right=False) + 1).clamp(max=I).tolist() + [I]
        return np.maximum.accumulate(np.asarray(cuts))

    def bridge_flags(self, S):
        """B1: has the item a neighbour of the other domain?  From the rows S holds (an item-sharded rank: its share; the
        flags of the other rows come out 0 and are all-reduced by the caller)."""
        I = self.R.n_items
        bb = self._zeros(max(I, 1), torch.uint8)
        check(lib.xmap_bridge_flags(_stream(self.dev), C.byref(S.c), vp(self.R.prefix_cls), vp(bb)))
        return bb

    def ext_knn(self, S, top_k, comm=None, rows=None, bb=None):
        """local part: the classified top-k lists of this rank's share of the rows (all rows without comm; rows = (lo, hi):
        that share -- S then holds only those rows, and bb must be the complete bridge flags: a neighbour's class is read)"""
        I = self.R.n_items
        if rows is None and comm is not None and comm.world > 1 and I > 0:
            w = comm.world
            tgt = (S.row_ptr[I].double() * torch.arange(1, w, dtype=torch.float64, device=self.dev) / w).long()
            cuts = [0] + torch.searchsorted(S.row_ptr[:I + 1].contiguous(), tgt).clamp(max=I).tolist() + [I]
            cuts = np.maximum.accumulate(np.asarray(cuts))
            rows = (int(cuts[comm.rank]), int(cuts[comm.rank + 1]))
        E = self.knn(S, top_k, bb=bb, rows=rows)
        E.rows = rows
        E.row_share = rows          # (kept for the reverse lists of a sharded step)
        ok = C.c_int32(1)
        check(lib.xmap_edge_ranges(_stream(self.dev), C.byref(S.c), C.byref(ok)))
        E.fast_div = int(ok.value)       # the bare division sequence of k_paths4 is the division for these edge values
        return E

    def ext_gather(self, E, comm):
        """collectives only: the ranks' shares of the knn tables, all-gathered"""
        rows = getattr(E, "rows", None)
        if rows is None:
            return E
        I = self.R.n_items
        lo, hi = rows
        with self.timed("knn_gather"):
            k = E.k
            fd = torch.tensor([E.fast_div], dtype=torch.int64, device=self.dev)      # (every rank checked its own rows' edges)
            comm.all_reduce(fd, "min")
            E.fast_div = int(fd.item())
            cls, kcnt, kcol, kval = comm.all_gather_multi([E.cls[lo:hi], E.kcnt[lo:hi].reshape(-1), E.kcol[lo:hi].reshape(-1),
                                                           E.kval[lo:hi].reshape(-1)])       # (one exchange for the four tables)
            E.cls[:I] = cls
            E.kcnt[:I] = kcnt.view(I, 2)
            E.kcol[:I] = kcol.view(I, 2, k)
            E.kval[:I] = kval.view(I, 2, k, 3)
        E.rows = None
        return E

    def ext_thresholds(self, E):
        I = self.R.n_items
        E.thr = self._empty(max(I, 1) * 4, torch.float64)      # last entry of every list, 16 B each
        check(lib.xmap_knn_thresholds(_stream(self.dev), i32(I), E.k, vp(E.kcnt), vp(E.kcol), vp(E.kval), vp(E.thr)))
        E.long_rows = self._empty(max(I, 1) + 1, torch.int32)

    def ext_reverse(self, S, E):
        """local part: list thresholds and the three reverse adjacencies from the complete knn tables (item-sharded ranks
        run the three in row shares: xmap.engine.sharded._stage_b)"""
        with self.timed("reverse"):
            self.ext_thresholds(E)
            if getattr(E, "thr", None) is not None and os.environ.get("XMAP_REV_SEPARATE") != "1":
                st_att, st_rnn = self.reverse_count_pair(S, E, None)      # attach + rnn: one count pass for both
                self.reverse_fill(st_att)
                self.reverse_fill(st_rnn)
                E.att, E.rnn = st_att.out, st_rnn.out
            else:
                E.att = self._reverse(S, E, 0, None)
                E.rnn = self._reverse(S, E, 2, None)
            E.src = self._reverse(S, E, 1, E.att[0])
        return E

    def ext_tables_from_knn(self, top_k, cls, kcnt, kcol, kval):
        """The same tables from classified top-k lists alone (host arrays: cls [I], kcnt [I][2], kcol [I][2][k],
        kval [I][2][k][3]) -- what ExtendSim.sim_extend is handed by a caller that ran find_knn_items /
        extract_siminfo itself (reference core/extender.py:46-81,171-178).  The reverse lists are small (<= 2 k I
        entries): built on the host, in list order."""
        R = self.R
        I, k = R.n_items, int(top_k)
        dev = self.dev
        flags = R.flags[:I].cpu().numpy()
        cls = np.ascontiguousarray(cls, np.uint8)
        kcnt = np.ascontiguousarray(kcnt, np.int32).reshape(I, 2)
        kcol = np.ascontiguousarray(kcol, np.int32).reshape(I, 2, k)
        kval = np.ascontiguousarray(kval, np.float64).reshape(I, 2, k, 3)

        def entries(rows, lst):
            """(row item, position) of the valid entries of list `lst` of the given rows"""
            c = kcnt[rows, lst]
            r = np.repeat(rows, c)
            q = np.arange(int(c.sum())) - np.repeat(np.cumsum(c) - c, c)
            return r, q

        def csr(keys, idx, vals, flag=None):
            o = np.argsort(keys, kind="stable")
            ptr = np.zeros(I + 1, np.int64)
            np.cumsum(np.bincount(keys, minlength=I), out=ptr[1:])
            t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a if len(a) else np.zeros((1,) + a.shape[1:]), dt)).to(dev)
            return (torch.from_numpy(ptr).to(dev), t(idx[o], np.int32), t(vals[o], np.float64),
                    t((flag[o] if flag is not None else np.zeros(len(o))), np.uint8), len(o))
        nb = np.nonzero(cls == 2)[0]
        # attach(b) = [x : x non-bridge record, b in NB_BB(x)];  rnn(y) = [x : y in NB_NN(x)]
        x0, q0 = entries(nb, 0)
        att = csr(kcol[x0, 0, q0].astype(np.int64), x0, kval[x0, 0, q0])
        x1, q1 = entries(nb, 1)
        rnn = csr(kcol[x1, 1, q1].astype(np.int64), x1, kval[x1, 1, q1])
        natt = np.diff(att[0].cpu().numpy())
        # src(t) = [s : s bridge, "S:" in s, attach(s) not empty, t in the lists of s, "T:" in t]; joint iff (t, s) is in TGT too
        bs = np.nonzero((cls == 1) & ((flags & 1) != 0) & (natt > 0))[0]
        s_, t_, v_ = [], [], []
        for lst in (0, 1):
            r, q = entries(bs, lst)
            s_.append(r); t_.append(kcol[r, lst, q]); v_.append(kval[r, lst, q])
        s_ = np.concatenate(s_) if s_ else np.zeros(0, np.int64)
        t_ = np.concatenate(t_).astype(np.int64) if t_ else np.zeros(0, np.int64)
        v_ = np.concatenate(v_) if v_ else np.zeros((0, 3))
        keep = (flags[t_] & 2) != 0
        s_, t_, v_ = s_[keep], t_[keep], v_[keep]
        # TGT: t bridge, "T:" in t, attach(t) not empty, s in the lists of t
        tgt_ok = (cls[t_] == 1) & (natt[t_] > 0)
        in_t = np.zeros(len(t_), bool)
        for lst in (0, 1):
            m = (np.arange(k)[None, :] < kcnt[t_, lst][:, None]) & (kcol[t_, lst, :] == s_[:, None])
            in_t |= m.any(axis=1)
        src = csr(t_, s_, v_, (tgt_ok & in_t).astype(np.uint8))
        E = ExtResult()
        E.k = k
        t = lambda a: torch.from_numpy(a).to(dev)
        E.cls, E.kcnt, E.kcol, E.kval = t(cls), t(kcnt), t(kcol), t(kval)
        E.bb = t((cls == 1).astype(np.uint8))
        E.att, E.src, E.rnn = att, src, rnn
        valid = np.arange(k)[None, None, :] < kcnt[:, :, None]
        mu, smv = kval[..., 1][valid], np.abs(kval[..., 0][valid] * kval[..., 1][valid])
        E.fast_div = int(bool(np.all(mu >= 1.0) and np.all((smv == 0) | ((smv > 2.0 ** -400) & (smv < 2.0 ** 400)))))
        return E

    def extend(self, S, top_k, full=False, start_range=None, n_slots=5120, xs_cap=None, chunk=None,
               start_split=None, algo="cols", comm=None):
        """extender_pipeline: knn tables, reverse adjacencies, streamed path enumeration."""
        return self.extend_tables(self.ext_tables(S, top_k, comm), full, start_range, n_slots, xs_cap, chunk, start_split, algo)

    def extend_lists(self, E):
        """the full X-Sim lists of a pass that kept the candidate arrays only (lazy extended_simRDD): the enumeration is
        run once more with the list buffers sized exactly from n_cand"""
        if getattr(E, "xs_end", None) is None:
            total = int(E.n_cand.sum().item())
            self.extend_tables(E, True, E.start_range, xs_cap=max(total, 1), algo=E.algo)
        return E

    def extend_tables(self, E, full=False, start_range=None, n_slots=5120, xs_cap=None, chunk=None,
                      start_split=None, algo="cols"):
        """B5c-B6 on prepared tables: work units, middle lists, path enumeration, fused top-XMAP_TOPC"""
        R = self.R
        I = R.n_items
        st = _stream(self.dev)
        E.algo = algo
        U = self.path_units(E, start_range, chunk, start_split=start_split)
        # The path count is the reference's algorithm, not the engine's: every joint (t, s) multiplies the attach lists of
        # both ends (core/extender.py:142-169), so a thin bridge set with long lists and a long k explode it (the S1 shape
        # of the reference's report has 2.9e10 paths at its own k = 10 and 5.8e13 at k = 50).  Refuse up front what would
        # take hours, instead of dying in an allocation.
        max_paths = float(os.environ.get("XMAP_MAX_PATHS", "5e12"))
        if U.total_all > max_paths:
            raise abi.XmapError(abi.ERR_CAPACITY, "the extension has %.3g paths at top_k = %d (limit XMAP_MAX_PATHS = %.3g): "
                                "use a shorter list" % (U.total_all, E.k, max_paths))
        M = getattr(E, "mid", None)
        if M is None:
            M = self.mid_lists(E) if algo in ("mid", "cols") else None
        if algo == "cols" and M is None:
            algo = "enum"           # no non-bridge records (nothing joint) or the lists do not fit: per-path enumeration
        E.units = U
        E.mid = M
        if algo == "cols":
            return self._extend_cols(E, U, M, full, xs_cap, start_range, n_slots)
        # one private accumulator row (36 B per item) per resident wave: 5 waves per SIMD = 5120 rows on 256 CUs
        slot_budget = int(float(os.environ.get("XMAP_SLOT_BUDGET_GB", "100")) * (1 << 30))
        n_slots = int(os.environ.get("XMAP_N_SLOTS", n_slots))
        n_slots = int(max(4, min(n_slots, slot_budget // (36 * max(I, 1)), max(U.n_units, 4))))
        acc = self._zero_scratch("acc", n_slots * max(I, 1) * 4, torch.float64)
        touched = self._empty(n_slots * max(I, 1), torch.int32)
        hacc = self._zero_scratch("hacc", max(U.n_rows, 1) * max(I, 1) * 4, torch.float64) if U.n_rows else None
        htouched = self._empty(max(U.n_rows, 1) * max(I, 1), torch.int32) if U.n_rows else None
        E.n_cand = self._zeros(max(I, 1), torch.int32)
        E.top_end = torch.full((max(I, 1), abi.TOPC), -1, dtype=torch.int32, device=self.dev)
        E.top_val = self._zeros((max(I, 1), abi.TOPC), torch.float64)
        d_cnt = self._zeros(4, torch.int64)
        h_cnt = (C.c_int64 * 4)()
        cap = 0
        if full:
            cap = int(xs_cap) if xs_cap else 1 << 22
        while True:
            xs_off = self._zeros(max(I, 1), torch.int64) if cap else None
            xs_end = self._empty(max(cap, 1), torch.int32) if cap else None
            xs_val = self._empty(max(cap, 1), torch.float64) if cap else None
            args = (st, i32(I), E.k, vp(E.cls), vp(E.kcnt), vp(E.kcol), vp(E.kval), vp(R.flags),
                    vp(E.att[0]), vp(E.att[1]), vp(E.att[2]),
                    vp(E.src[0]), vp(E.src[1]), vp(E.src[2]), vp(E.src[3]),
                    vp(E.rnn[0]), vp(E.rnn[1]), vp(E.rnn[2]),
                    i32(U.n_units), vp(U.unit_start), vp(U.unit_c), vp(U.unit_G), vp(U.unit_row), vp(U.unit_nt),
                    i32(U.n_heavy), vp(U.heavy_unit0),
                    i32(n_slots), vp(acc), vp(touched), vp(hacc), vp(htouched),
                    vp(E.n_cand), vp(E.top_end), vp(E.top_val),
                    i64(cap), vp(xs_off), vp(xs_end), vp(xs_val), vp(d_cnt), h_cnt)
            with self.timed("paths"):
                if M is not None:
                    rc = abi.xlib().xmap_extend_paths2(*args, vp(M.nb_id), vp(M.nb_list), i32(M.n_nb), vp(M.midX), vp(M.dir),
                                                vp(M.dir_ptr), vp(M.ng))
                else:
                    rc = lib.xmap_extend_paths(*args)
            if rc == abi.ERR_CAPACITY:
                cap = int(h_cnt[0])
                continue
            if rc != 0:
                self._drop_scratch("acc", "hacc")
            (abi.xcheck if M is not None else check)(rc)
            break
        E.n_out, E.n_paths = int(h_cnt[0]), int(h_cnt[1])
        E.xs_off, E.xs_end, E.xs_val = xs_off, xs_end, xs_val
        E.start_range = (0, I) if start_range is None else tuple(start_range)
        return E

    # ------------------------------------------------------------------ dense item-factor variant
    def dense_topk(self, F_t, F_s, top_k):
        """Row-wise top-k of cosine(F_t[i], F_s[j]) by (|sim| desc, j asc) on the fp32 matrix cores.
        F_t [n_t, K], F_s [n_s, K]: host or device float32.  Returns (idx int32 [n_t, k], val float32 [n_t, k])."""
        st = _stream(self.dev)
        to = lambda a: (a if torch.is_tensor(a) else torch.from_numpy(np.ascontiguousarray(a, np.float32))).to(
            self.dev, torch.float32).contiguous()
        Ft, Fs = to(F_t), to(F_s)
        n_t, K = Ft.shape
        n_s = Fs.shape[0]
        Fnt, Fns = torch.empty_like(Ft), torch.empty_like(Fs)
        with self.timed("dense_normalize"):
            check(lib.xmap_dense_normalize(st, i32(n_t), i32(K), vp(Ft), vp(Fnt)))
            check(lib.xmap_dense_normalize(st, i32(n_s), i32(K), vp(Fs), vp(Fns)))
        idx = self._empty((max(n_t, 1), top_k), torch.int32)
        val = self._empty((max(n_t, 1), top_k), torch.float32)
        npc = C.c_int32(1)
        check(lib.xmap_dense_layout(i32(n_t), i32(n_s), C.byref(npc)))
        n_pieces = int(npc.value)
        pidx = self._empty((max(n_t, 1), n_pieces, top_k), torch.int32) if n_pieces > 1 else None
        pval = self._empty((max(n_t, 1), n_pieces, top_k), torch.float32) if n_pieces > 1 else None
        with self.timed("dense_topk"):
            check(lib.xmap_dense_topk(st, i32(n_t), i32(n_s), i32(K), vp(Fnt), vp(Fns), i32(top_k), i32(n_pieces),
                                      vp(pidx), vp(pval), vp(idx), vp(val)))
        return idx[:n_t], val[:n_t]

    def dense_extend(self, F, top_k):
        """Dense replacement of stages A+B (BASELINE configs[4]): F [I, K] item factors; target-side items ("T:" in
        iid) are the starts, source-side items ("S:" in iid) the candidates.  Returns an ExtResult with the same
        candidate arrays extend() produces, so select() / alterego() run unchanged."""
        R = self.R
        I = R.n_items
        F = (F if torch.is_tensor(F) else torch.from_numpy(np.ascontiguousarray(F, np.float32))).to(self.dev)
        tgt = torch.nonzero((R.flags[:I] & 2) != 0).flatten()
        src = torch.nonzero((R.flags[:I] & 1) != 0).flatten()
        idx, val = self.dense_topk(F[tgt], F[src], top_k)
        E = ExtResult()
        E.k = top_k
        E.dense_idx, E.dense_val, E.tgt, E.src = idx, val, tgt, src
        E.n_cand = self._zeros(max(I, 1), torch.int32)
        E.top_end = torch.full((max(I, 1), abi.TOPC), -1, dtype=torch.int32, device=self.dev)
        E.top_val = self._zeros((max(I, 1), abi.TOPC), torch.float64)
        c = min(abi.TOPC, top_k)
        valid = idx >= 0
        E.n_cand[tgt] = valid.sum(dim=1).to(torch.int32)
        ends = torch.where(valid[:, :c], src[idx[:, :c].clamp(min=0).long()].to(torch.int32),
                           torch.full_like(idx[:, :c], -1))
        E.top_end[tgt, :c] = ends
        E.top_val[tgt, :c] = val[:, :c].double()
        return E

    # ------------------------------------------------------------------ stage C
    def select(self, E, private, picks=None):
        R = self.R
        st = _stream(self.dev)
        I = R.n_items
        n_top = self._out(max(I, 1), torch.int32, I > 0)      # written for every start / filled by the library
        choice = self._out(max(I, 1), torch.int32, I > 0)
        mp = self._out(max(I, 1), torch.int32, I > 0)
        pk = None
        if picks is not None:
            pk = torch.from_numpy(np.ascontiguousarray(picks, np.int32)).to(self.dev)
        with self.timed("c_select"):
            check(lib.xmap_select_map(st, i32(I), 1 if private else 0, vp(E.n_cand), vp(E.top_end), vp(pk),
                                      vp(n_top), vp(choice), vp(mp)))
        return n_top, choice, mp

    def alterego(self, mp, users=None):
        """build_alterEgo (core/generator.py:113-157) over all users, or over the users [users[0], users[1])"""
        R = self.R
        st = _stream(self.dev)
        U = R.n_users
        Rc = R.c
        if users is not None:       # a share of the users (an item-sharded rank's part of stage C): rows carry GLOBAL user indices
            u_lo, u_hi = int(users[0]), int(users[1])
            U = u_hi - u_lo
            Rc = abi.Ratings.from_buffer_copy(R.c)
            Rc.n_users = U
            Rc.user_ptr = R.user_ptr.data_ptr() + 8 * u_lo
        cnt_t = self._empty(max(U, 1), torch.int32)
        cnt_m = self._empty(max(U, 1), torch.int32)
        if U == 0:
            cnt_t.zero_(); cnt_m.zero_()
        d_prof = self._zeros(64, torch.int64)            # sharded counter of the users with output rows
        with self.timed("c_count"):
            check(lib.xmap_alterego_count(st, C.byref(Rc), vp(mp), vp(cnt_t), vp(cnt_m), vp(d_prof)))
            off_t = self._empty(U + 1, torch.int64)
            off_m = self._empty(U + 1, torch.int64)
            check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(cnt_t), vp(off_t), i64(U), None))
            check(lib.xmap_exclusive_scan_i32_to_i64(st, vp(cnt_m), vp(off_m), i64(U), None))
            # the three totals in ONE synchronisation (the scans leave theirs in out[U])
            h = self._pinned3()
            h[0:1].copy_(off_t[U:U + 1], non_blocking=True)
            h[1:2].copy_(off_m[U:U + 1], non_blocking=True)
            h[2:66].copy_(d_prof, non_blocking=True)
            torch.cuda.current_stream(self.dev).synchronize()
            nt, nm, n_prof = int(h[0]), int(h[1]), int(h[2:66].sum())
        n = nt + nm
        G = GenResult()
        G.user = self._empty(max(n, 1), torch.int32)
        G.item = self._empty(max(n, 1), torch.int32)
        G.rating = self._empty(max(n, 1), torch.float64)      # pass-through ratings and np.mean of the merged ones (fp64)
        G.time = self._empty(max(n, 1), torch.int64)
        with self.timed("c_fill"):
            check(lib.xmap_alterego_fill(st, C.byref(Rc), vp(mp), vp(off_t), vp(off_m), i64(nt),
                                         vp(G.user), vp(G.item), vp(G.rating), vp(G.time)))
            if users is not None and u_lo:
                G.user += u_lo
        G.n_rows, G.n_target_rows = n, nt
        G.cnt_t, G.cnt_m = cnt_t, cnt_m
        G.n_profiles = n_prof
        G.user, G.item, G.rating, G.time = G.user[:n], G.item[:n], G.rating[:n], G.time[:n]
        return G

    def _pinned3(self):
        h = getattr(self, "_h3", None)
        if h is None:
            h = self._h3 = torch.empty(66, dtype=torch.int64, pin_memory=True)
        return h

    def n_profiles(self, G):
        """distinct users present in the AlterEgo output (profiles/s numerator)."""
        n = getattr(G, "n_profiles", None)      # counted by the count pass of alterego()
        if n is not None:
            return int(n)
        U = self.R.n_users
        return int(((G.cnt_t[:U] + G.cnt_m[:U]) > 0).sum().item())


def draw_picks(n_top):
    """cross_nonprivate_mapping's draw (core/generator.py:110): one np.random.randint(0, len(top)-1)
    per start item in ascending start order from the GLOBAL NumPy RNG (the vectorised call consumes
    the stream exactly like the reference's sequential scalar calls).  Singleton candidate lists raise
    ValueError like the reference."""
    n_top = np.asarray(n_top)
    starts = np.nonzero(n_top)[0]
    picks = np.zeros(len(n_top), np.int32)
    if len(starts):
        high = n_top[starts].astype(np.int64) - 1
        if (high <= 0).any():
            raise ValueError("low >= high")
        picks[starts] = np.random.randint(0, high)
    return picks
